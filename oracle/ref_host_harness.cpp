/*
 * oracle/ref_host_harness.cpp -- C exports over the reference's own HOST builder and atmosphere, for
 * oracle/_ref/libref_traverse.so.  TEST INFRASTRUCTURE (see orc.h); built in the authoring container only.
 *
 * This translation unit contains NO reference code.  oracle/Makefile compiles, unmodified and where they lie under
 * /root/reference/PathTracer:
 *     bvh.cpp    (BVH::BVH, computeBucket, recursiveBuild, initLeaf / initInterior   bvh.cpp:3-225)
 *     Bbox.cpp   (Union                                                               Bbox.cpp:3-14)
 *     sunsky.cu  (RayleighPhase, totalMie, hgPhase, SunIntensity, fromSpherical, sun, sky, sunsky, ortho,
 *                 getConeSample                                                       sunsky.cu:10-185)
 * as C++ (`-x c++ -include math.h` for sunsky.cu: the system header that puts the float overloads of pow / cos / exp /
 * acos into the global namespace, which is what nvcc does), against the vendored glm / glad / GLFW / assimp HEADERS and
 * the real CUDA runtime headers of this image.  The one thing their `stdafx.h` cannot find on a case-sensitive file
 * system is "BVH.h" (stdafx.h:42; the file is bvh.h): the Makefile makes `_ref/case/BVH.h` a SYMLINK to the reference's
 * own bvh.h -- no header is written, no content substituted.
 *
 * What this file adds is only the glue a C caller needs:
 *   ref_bvh_build      constructs the reference's `BVH` over the caller's arrays and copies nodes / reordered
 *                      primitives out (what Scene.cpp:53-65 does with them)
 *   ref_sun_setup      stores the three device globals the way launch_kernels does (kernel.cu:683-684, 704-709);
 *                      kernel.cu itself needs nvcc, so those two assignments are restated here, from the reference's
 *                      own `fromSpherical`, `sunSize`, `pi` and the vendored glm::normalize
 *   RandomFloat2       sunsky.cu:168 declares it and kernel.cu:23-37 defines it (nvcc only): restated below so that
 *                      getConeSample links; the xorshift stream itself is pinned by tests/golden/kat.json, so the
 *                      cone arithmetic is pinned GIVEN that stream
 *   ref_sun / ref_sky / ref_sunsky / ref_cone_sample / ref_sun_helpers   array wrappers
 */
#include "stdafx.h"
#include "sunsky.cuh"

#include <sstream>

/* kernel.cu:23-28, 34-37 (see the header comment) */
static unsigned int harness_random_int(unsigned int& seed) {
	seed ^= seed << 13;
	seed ^= seed >> 17;
	seed ^= seed << 5;
	return seed;
}
float RandomFloat2(unsigned int& seed) { return (harness_random_int(seed) >> 16) / 65535.0f; }

extern "C" {

/* prims: n x 40 B Triangle records, reordered in place (bvh.cpp:24); bboxes: n x 24 B; nodes_out: room for 2n-1 x 32 B.
 * algo = (int)PartitionAlgorithm (bvh.h:44-46: 0 Middle, 1 EqualCounts, 2 SAH).  Returns nNodes. */
int ref_bvh_build(void* prims, int n, const void* bboxes, void* nodes_out, int algo) {
	std::vector<Triangle> P((size_t)n);
	std::vector<BBox> B((size_t)n);
	std::memcpy(P.data(), prims, (size_t)n * sizeof(Triangle));
	std::memcpy(B.data(), bboxes, (size_t)n * sizeof(BBox));
	std::ostringstream sink; /* the constructor prints its statistics (bvh.cpp:7, 27-42) */
	std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
	BVH bvh(P, B, (PartitionAlgorithm)algo);
	std::cout.rdbuf(old);
	if (n > 0) {
		std::memcpy(nodes_out, bvh.nodes.data(), (size_t)bvh.nNodes * sizeof(BVH::BVHNode));
		std::memcpy(prims, P.data(), (size_t)n * sizeof(Triangle));
	}
	return bvh.nNodes;
}

/* Bbox.cpp:3-14 over arrays of 24-byte boxes */
void ref_bbox_union(const void* a, const void* b, int n, void* out) {
	const BBox* A = (const BBox*)a;
	const BBox* Bb = (const BBox*)b;
	BBox* O = (BBox*)out;
	for (int i = 0; i < n; ++i)
		O[i] = Union(A[i], Bb[i]);
}

/* kernel.cu:683-684 and 704-709.  out8 = sunDirection.xyz, sunAngularDiameterCos, fromSpherical(p).xyz (unnormalised),
 * SunIntensity(dot(sunDirection, up)) (the sunE of sunsky.cu:38) */
void ref_sun_setup(const float* sun_position2, float* out8) {
	const glm::vec2 sun_position(sun_position2[0], sun_position2[1]);
	const float sun_angular = cos(sunSize * pi / 180.f);
	sunAngularDiameterCos = sun_angular;
	SunPos = sun_position;
	const glm::vec3 sph = fromSpherical((sun_position - glm::vec2(0.0, 0.5)) * glm::vec2(6.28f, 3.14f));
	const glm::vec3 sun_direction = glm::normalize(sph);
	sunDirection = sun_direction;
	out8[0] = sun_direction.x;
	out8[1] = sun_direction.y;
	out8[2] = sun_direction.z;
	out8[3] = sun_angular;
	out8[4] = sph.x;
	out8[5] = sph.y;
	out8[6] = sph.z;
	out8[7] = SunIntensity(dot(sunDirection, up));
}

/* which: 0 sun (sunsky.cu:32-74), 1 sky (76-114), 2 sunsky (116-161); dirs / out are n x float3; uses the globals ref_sun_setup stored */
int ref_atmosphere(int which, const float* dirs, int n, float* out) {
	for (int i = 0; i < n; ++i) {
		const glm::vec3 d(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
		glm::vec3 r;
		switch (which) {
		case 0: r = sun(d); break;
		case 1: r = sky(d); break;
		case 2: r = sunsky(d); break;
		default: return -1;
		}
		out[3 * i] = r.x;
		out[3 * i + 1] = r.y;
		out[3 * i + 2] = r.z;
	}
	return 0;
}

/* the scalar helpers (sunsky.cu:10-26): out4 per input = RayleighPhase(x), hgPhase(x, mieDirectionalG), SunIntensity(x), 0 */
void ref_sun_helpers(const float* x, int n, float* out4) {
	for (int i = 0; i < n; ++i) {
		out4[4 * i] = RayleighPhase(x[i]);
		out4[4 * i + 1] = hgPhase(x[i], mieDirectionalG);
		out4[4 * i + 2] = SunIntensity(x[i]);
		out4[4 * i + 3] = 0.0f;
	}
}

/* totalMie(primaryWavelengths, K, turbidity) * mieCoefficient (sunsky.cu:15-19, 44) */
void ref_mie_at_x(float* out3) {
	const glm::vec3 m = totalMie(primaryWavelengths, K, turbidity) * mieCoefficient;
	out3[0] = m.x;
	out3[1] = m.y;
	out3[2] = m.z;
}

/* getConeSample(sunDirection, 1 - sunAngularDiameterCos, seed) as kernel.cu:410-411 calls it, n times along one seed
 * stream (seeds_io[0] in, the state after the n-th call out); out is n x float3 */
void ref_cone_samples(unsigned int* seed_io, int n, float* out) {
	unsigned int seed = *seed_io;
	for (int i = 0; i < n; ++i) {
		const glm::vec3 r = getConeSample(sunDirection, 1.0f - sunAngularDiameterCos, seed);
		out[3 * i] = r.x;
		out[3 * i + 1] = r.y;
		out[3 * i + 2] = r.z;
	}
	*seed_io = seed;
}

} /* extern "C" */
