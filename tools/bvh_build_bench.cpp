#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../include/tyr_c.h" // g++ -O2 tools/bvh_build_bench.cpp -Ltyrant_amd/lib -ltyrant_hip -Wl,-rpath,$PWD/tyrant_amd/lib
int main(int argc, char** argv) {
	int side = argc > 1 ? atoi(argv[1]) : 1000;
	int n = side * side * 2;
	std::vector<tyr_triangle> t(n);
	unsigned s = 1;
	auto rnd = [&] { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (s & 0xffffff) / 16777216.0f; };
	for (int y = 0; y < side; ++y) for (int x = 0; x < side; ++x) for (int k = 0; k < 2; ++k) {
		tyr_triangle& q = t[(y * side + x) * 2 + k];
		float* f = (float*)&q;
		for (int i = 0; i < 10; ++i) f[i] = 0;
		float z = 10 * sinf(x * 0.01f) * cosf(y * 0.013f) + rnd();
		f[0] = x; f[1] = y; f[2] = z; f[3] = x + 1; f[4] = y + (k ? 1 : 0); f[5] = z + rnd(); f[6] = x + (k ? 0 : 1); f[7] = y + 1; f[8] = z + rnd();
	}
	std::vector<tyr_bbox> bb(n);
	tyr_triangle_bboxes(t.data(), n, bb.data());
	std::vector<tyr_bvh_node> nodes(2 * (size_t)n);
	for (int th : {1, 2, 4, 8}) {
		std::vector<tyr_triangle> p = t;
		tyr_set_build_threads(th);
		auto t0 = std::chrono::steady_clock::now();
		int nn = tyr_bvh_build(p.data(), n, bb.data(), nodes.data(), 2);
		double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		printf("n=%d threads=%d nodes=%d %.3f s\n", n, th, nn, dt);
	}
}
