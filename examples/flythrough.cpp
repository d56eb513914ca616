// examples/flythrough.cpp -- the reference's PERFORMANCE_TEST build (main.cpp:7, 115-117, 153-158;
// performance_measure.cpp:7-45) without the window: the camera is parked at three recorded views, T seconds each, one
// launch_kernels + std::swap per frame (main.cpp:166-169), and per view the average / minimum / maximum frame time goes
// to Performance.txt in the reference's layout.  Between the views the scripted input of the interactive build
// (Camera::handle_input, camera.cpp:3-44, as a pure function of a key / cursor record) walks the camera for a few
// frames, which -- like any camera change -- resets the accumulation (kernel.cu:702-718): the path count of a probe
// pixel is printed so the accumulate / reset behaviour is visible.
//
//   flythrough [device] [seconds_per_view = 10] [Performance.txt] [K = 0] [prefix = frame]
//
// K > 0 is the progressive display of the interactive build (main.cpp:164-203) in the form a headless node can have: every
// K-th frame what interop.blit() would put on the screen -- blit_onto_framebuffer's tone-mapped picture of the
// accumulation so far (kernel.cu:648-662) -- goes to <prefix>_<frame>.ppm, and at the end <prefix>_hud.txt holds what the
// "Performance" window of main.cpp:177-198 shows: the average over the last 120 frames as ImGui's IO.Framerate keeps it,
// the last 200 frame times that PlotHistogram draws (one per line, seconds, as main.cpp:179-183 collects them), camera
// position and angles, the sun position.
//
// Layout of the file, as performance_measure.cpp:27-31 writes it (quirk kept: the "Min ms" / "Max ms" lines hold
// SECONDS -- `delta` is printed unscaled there; the true milliseconds go to stdout):
//     Average ms: <1000 * mean delta>
//     Average fps: <1 / mean delta>
//     Min ms: <min delta, seconds>
//     Max ms: <max delta, seconds>
//     <blank line between views>
#define TYRANT_IMPLEMENTATION
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <numeric>
#include <string>
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

// a height field standing in for Data/castle.ply (absent from the reference checkout), under the reference's spheres
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

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
	const int device = argc > 1 ? std::atoi(argv[1]) : 0;
	const double seconds_per_view = argc > 2 ? std::atof(argv[2]) : 10.0; // performance_measure.cpp:24
	const char* out_path = argc > 3 ? argv[3] : "Performance.txt";         // performance_measure.cpp:4
	const unsigned progressive = argc > 4 ? static_cast<unsigned>(std::atoi(argv[4])) : 0u;
	const std::string prefix = argc > 5 ? argv[5] : "frame";
	const unsigned W = 1920, H = 1080, N = ray_queue_buffer_size;          // variables.h:9-10, 44

	// performance_measure.h:4-5
	const std::vector<vec3> test_positions = { { -0.119f, -26.116f, 32.537f }, { -52.741f, -44.67f, 109.04f }, { 74.65f, 2.77f, 17.336f } };
	const std::vector<vec2> test_angles = { { 12.576f, -0.518204f }, { -6470.568f, -0.818204f }, { -10218.468f, 0.081796f } };

	tyr_config cfg{};
	cfg.width = W;
	cfg.height = H;
	cfg.queue_size = N;
	cfg.device = device;
	cfg.nranks = 1;
	tyr_ctx* ctx = nullptr;
	TYR_CHECK(tyr_create(&ctx, &cfg));
	Scene scene;
	scene.Load(ctx, make_mesh(256));
	RayQueue *ray_buffer_work = nullptr, *ray_buffer_next = nullptr;
	ShadowQueue* shadow_queue_buffer = nullptr;
	vec4* blit_buffer = nullptr;
	TYR_CHECK(hipSetDevice(device));
	TYR_CHECK(hipMalloc(reinterpret_cast<void**>(&blit_buffer), sizeof(vec4) * W * H));

	std::ofstream file(out_path);
	const size_t probe = static_cast<size_t>(H / 2) * W + W / 2;
	auto probe_paths = [&]() {
		vec4 px{};
		TYR_CHECK(hipMemcpy(&px, blit_buffer + probe, sizeof px, hipMemcpyDeviceToHost));
		return px.w;
	};

	// the progressive display (K > 0): the picture interop.blit() would show, every K-th frame
	vec4* screen = nullptr;
	std::vector<float> screen_host;
	std::vector<float> frame_times; // main.cpp:177-183: the last 200 deltas
	unsigned long long frame_no = 0;
	unsigned pictures = 0;
	if (progressive) {
		TYR_CHECK(hipMalloc(reinterpret_cast<void**>(&screen), sizeof(vec4) * W * H));
		screen_host.resize(static_cast<size_t>(W) * H * 4);
	}
	auto hud_frame = [&](double delta) {
		frame_times.push_back(static_cast<float>(delta)); // main.cpp:179
		if (frame_times.size() > 200)
			frame_times.erase(frame_times.begin());        // main.cpp:181-183
		++frame_no;
		if (progressive && frame_no % progressive == 0) {
			TYR_CHECK(tyr_resolve(ctx, screen)); // kernel.cu:729-731 (the surface write) + interop.blit()
			TYR_CHECK(hipMemcpy(screen_host.data(), screen, sizeof(vec4) * W * H, hipMemcpyDeviceToHost));
			const std::string name = prefix + "_" + std::to_string(frame_no) + ".ppm";
			TYR_CHECK(tyr_write_ppm(name.c_str(), screen_host.data(), W, H));
			++pictures;
		}
	};

	double previous_time = now_s();
	for (size_t current_test = 0; current_test < test_positions.size(); ++current_test) {
		if (current_test > 0) {
			// the interactive build's input path for a few frames: W + LEFT_SHIFT held, the cursor 40 px right of the centre
			tyr_input_state in{};
			in.key_w = in.key_left_shift = 1;
			in.window_w = static_cast<int32_t>(W);
			in.window_h = static_cast<int32_t>(H);
			in.cursor_x = W * 0.5 + 40.0;
			in.cursor_y = H * 0.5;
			for (int f = 0; f < 8; ++f) {
				const double delta = now_s() - previous_time; // main.cpp:140-141
				previous_time = now_s();
				camera.handle_input(in, delta); // main.cpp:164
				camera.update();                // main.cpp:166
				TYR_CHECK(launch_kernels(nullptr, blit_buffer, scene.gpuScene, ray_buffer_work, ray_buffer_next, shadow_queue_buffer));
				std::swap(ray_buffer_work, ray_buffer_next);
				hud_frame(delta);
			}
			std::printf("walked 8 frames under scripted input: every frame moved the camera, probe pixel holds %.0f finished paths (reset each frame)\n", probe_paths());
		}
		// performance_measure.cpp:8-45 for one view
		double last_time = now_s(), delta_min = std::numeric_limits<double>::max(), delta_max = 0;
		std::vector<float> times;
		unsigned frames = 0;
		for (;;) {
			const double delta = now_s() - previous_time; // main.cpp:140-141
			previous_time = now_s();
			times.push_back(static_cast<float>(delta));
			camera.position = test_positions[current_test];
			camera.horizontal_angle = test_angles[current_test].x;
			camera.vertical_angle = test_angles[current_test].y;
			delta_min = std::min(delta_min, delta);
			delta_max = std::max(delta_max, delta);
			if (now_s() - last_time > seconds_per_view)
				break;
			camera.update(); // main.cpp:166
			TYR_CHECK(launch_kernels(nullptr, blit_buffer, scene.gpuScene, ray_buffer_work, ray_buffer_next, shadow_queue_buffer));
			std::swap(ray_buffer_work, ray_buffer_next); // main.cpp:169
			hud_frame(delta);
			++frames;
		}
		const double average_delta = std::accumulate(times.begin(), times.end(), 0.f) / times.size();
		file << "Average ms: " << average_delta * 1000.0 << "\n";
		file << "Average fps: " << 1.0 / average_delta << "\n";
		file << "Min ms: " << delta_min << "\n";
		file << "Max ms: " << delta_max << "\n";
		if (current_test + 1 < test_positions.size())
			file << "\n";
		tyr_counters k;
		TYR_CHECK(tyr_get_counters(ctx, &k));
		std::printf("view %zu: %u frames in %.2f s: average %.4f ms (%.1f fps), min %.4f ms, max %.4f ms; the camera stood still, probe pixel accumulated %.0f finished paths; frame counter %u\n", current_test + 1,
			frames, seconds_per_view, average_delta * 1e3, 1.0 / average_delta, delta_min * 1e3, delta_max * 1e3, probe_paths(), k.frame);
	}
	file.close();
	if (progressive) {
		// the "Performance" window, main.cpp:185-196
		const size_t m = std::min<size_t>(frame_times.size(), 120); // ImGui's IO.Framerate: the average over its last 120 frames
		double sum = 0.0;
		for (size_t i = frame_times.size() - m; i < frame_times.size(); ++i)
			sum += frame_times[i];
		const double framerate = m && sum > 0.0 ? m / sum : 0.0;
		std::ofstream hud(prefix + "_hud.txt");
		char line[256];
		std::snprintf(line, sizeof line, "Application average %.3f ms/frame (%.1f FPS)\n", framerate > 0.0 ? 1000.0 / framerate : 0.0, framerate);
		hud << line << "Frametimes (" << frame_times.size() << ")\n";
		for (float t : frame_times)
			hud << t << "\n";
		std::snprintf(line, sizeof line, "X: %f, Y: %f, Z: %f\nHor: %f, Vert: %f\nSun X: %f Y: %f\n", camera.position.x, camera.position.y, camera.position.z, camera.horizontal_angle, camera.vertical_angle, sun_position.x, sun_position.y);
		hud << line;
		std::printf("progressive display: %u pictures (every %u-th of %llu frames) as %s_<frame>.ppm, HUD text in %s_hud.txt\n", pictures, progressive, frame_no, prefix.c_str(), prefix.c_str());
		(void)hipFree(screen);
	}
	tyr_counters k;
	TYR_CHECK(tyr_get_counters(ctx, &k));
	std::printf("wrote %s; %.1f M rays traced, device_error %u\n", out_path, (k.total_extend_rays + k.total_shadow_rays) / 1e6, k.device_error);
	const int rc = k.device_error ? 1 : 0;
	TYR_CHECK(tyr_destroy(ctx));
	(void)hipFree(blit_buffer);
	return rc;
}
