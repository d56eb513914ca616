// tools/bvh_tsan_main.cpp -- driver of `make -C tyrant_amd/csrc tsan`: the task-parallel host BVH builder
// (tyrant_amd/csrc/host/bvh_build.cpp) under ThreadSanitizer.  CPU build only (sanitizers are not
// available on the GPU pool).  Builds a seeded soup serially and with 2 / 4 / 8 threads, several times each so that
// the FIFO's hand-offs interleave differently, and compares node and primitive bytes with the serial build.
// Exit code: 0 = identical and no report; 1 = bytes differ; 66 = ThreadSanitizer found a race (TSAN_OPTIONS).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/tyr_c.h"

namespace tyr {
int bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo);
void triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out);
void set_build_threads(int threads);
} // namespace tyr

int main(int argc, char** argv) {
	const int n = argc > 1 ? std::atoi(argv[1]) : 120000; // above the task grain (>= 4096 primitives per task, 8 tasks per thread)
	std::vector<tyr_triangle> tris(n);
	unsigned s = 12345u;
	auto rnd = [&] {
		s ^= s << 13;
		s ^= s >> 17;
		s ^= s << 5;
		return (s & 0xffffffu) / 16777216.0f;
	};
	for (auto& t : tris) {
		std::memset(&t, 0, sizeof t);
		for (int k = 0; k < 3; ++k) {
			t.vert[k] = 100.0f * rnd() - 50.0f;
			t.e1[k] = 3.0f * rnd() - 1.5f;
			t.e2[k] = 3.0f * rnd() - 1.5f;
		}
	}
	std::vector<tyr_bbox> bb(n);
	tyr::triangle_bboxes(tris.data(), n, bb.data());
	std::vector<tyr_bvh_node> ref(2 * static_cast<size_t>(n)), nodes(2 * static_cast<size_t>(n));
	std::vector<tyr_triangle> refPrims = tris;
	tyr::set_build_threads(1);
	const int nRef = tyr::bvh_build(refPrims.data(), n, bb.data(), ref.data(), 2);
	if (nRef <= 0)
		return 1;
	for (int threads : { 2, 4, 8 }) {
		for (int rep = 0; rep < 3; ++rep) {
			std::vector<tyr_triangle> p = tris;
			std::memset(nodes.data(), 0, nodes.size() * sizeof(tyr_bvh_node));
			tyr::set_build_threads(threads);
			const int nn = tyr::bvh_build(p.data(), n, bb.data(), nodes.data(), 2);
			if (nn != nRef || std::memcmp(nodes.data(), ref.data(), static_cast<size_t>(nn) * sizeof(tyr_bvh_node)) != 0 || std::memcmp(p.data(), refPrims.data(), p.size() * sizeof(tyr_triangle)) != 0) {
				std::printf("threads %d rep %d: DIFFERS from the serial build (%d vs %d nodes)\n", threads, rep, nn, nRef);
				return 1;
			}
		}
		std::printf("threads %d: %d nodes, byte-identical to the serial build (3 runs)\n", threads, nRef);
	}
	std::printf("bvh_tsan: ok (%d triangles)\n", n);
	return 0;
}
