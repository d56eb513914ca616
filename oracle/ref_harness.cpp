/*
 * oracle/ref_harness.cpp -- build recipe for oracle/_ref/libref_traverse.so.
 * TEST INFRASTRUCTURE (see orc.h); built in the authoring container only, the
 * resulting .so travels to the GPU box, the reference sources never do.
 *
 * This translation unit contains NO reference code.  It #includes, where they
 * lie under /root/reference/PathTracer, the reference headers whose functions
 * are header-only (variables.h, loader.h, Bbox.h, bvh.h, sunsky.cuh), together
 * with the vendored glm and the real CUDA runtime headers present in this image
 * (which define __host__/__device__ for host compilers), and exports the
 * reference functions through a C ABI so the CPU restatement can be pinned to
 * them bit-for-bit:
 *   CachedBVH::intersect / intersect_debug / intersectSimple   bvh.h:118-256
 *   BBox::intersect, addVertex, surfaceArea, largestExtent      Bbox.h:8-62
 *   Triangle::intersect                                         loader.h:21-46
 *   struct layouts and constants                                variables.h, bvh.h, sunsky.cuh
 *
 * The reference's own bvh.cpp, Bbox.cpp and sunsky.cu are compiled, unmodified, by the
 * second harness beside this one (ref_host_harness.cpp, `make ref`: their stdafx.h finds
 * "BVH.h" through a symlink to bvh.h the Makefile creates).  Not buildable here
 * (DESIGN.md "Oracle"): kernel.cu (nvcc: <<< >>>, surface<>, atomicAdd) and Scene.cpp /
 * static_mesh.cpp (assimp ships as a Windows-only .lib).
 */
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include <cuda_runtime.h>

#include "glm.hpp"

#include "variables.h"
#include "loader.h"
#include "Bbox.h"
#include "bvh.h"
#include "sunsky.cuh"

extern "C" {

/* sizeof / offsetof of the data contracts (SURVEY.md section 8 header) */
int ref_layout(int* out, int cap) {
	const int vals[] = {
		(int)sizeof(RayQueue), (int)offsetof(RayQueue, origin), (int)offsetof(RayQueue, direction), (int)offsetof(RayQueue, direct),
		(int)offsetof(RayQueue, distance), (int)offsetof(RayQueue, identifier), (int)offsetof(RayQueue, bounces), (int)offsetof(RayQueue, index),
		(int)offsetof(RayQueue, geometry_type), (int)offsetof(RayQueue, lastSpecular),
		(int)sizeof(ShadowQueue), (int)offsetof(ShadowQueue, origin), (int)offsetof(ShadowQueue, direction), (int)offsetof(ShadowQueue, color),
		(int)offsetof(ShadowQueue, buffer_index), (int)offsetof(ShadowQueue, closestDistance),
		(int)sizeof(Triangle), (int)offsetof(Triangle, vert), (int)offsetof(Triangle, e1), (int)offsetof(Triangle, e2), (int)offsetof(Triangle, materialType),
		(int)sizeof(BVH::BVHNode), (int)offsetof(BVH::BVHNode, bbox), (int)offsetof(BVH::BVHNode, primitiveOffset), (int)offsetof(BVH::BVHNode, secondChildOffset),
		(int)offsetof(BVH::BVHNode, primitiveCount), (int)offsetof(BVH::BVHNode, splitAxis),
		(int)sizeof(BBox),
	};
	const int n = (int)(sizeof(vals) / sizeof(vals[0]));
	for (int i = 0; i < n && i < cap; ++i)
		out[i] = vals[i];
	return n;
}

/* constants of variables.h and sunsky.cuh */
int ref_constants(double* out, int cap) {
	RayQueue rq{};
	ShadowQueue sq{};
	const double vals[] = {
		(double)pi, (double)inv_pi, (double)render_width, (double)render_height, (double)epsilon, (double)ray_queue_buffer_size,
		(double)(int)rq.geometry_type, (double)rq.lastSpecular, (double)sq.closestDistance,
		(double)sunSize, (double)cutoffAngle, (double)steepness, (double)SkyFactor, (double)turbidity, (double)mieCoefficient,
		(double)mieDirectionalG, (double)v, (double)rayleighZenithLength, (double)mieZenithLength, (double)sunIntensity,
		(double)primaryWavelengths.x, (double)primaryWavelengths.y, (double)primaryWavelengths.z,
		(double)(int)GeometryType::Sphere, (double)(int)GeometryType::Triangle,
	};
	const int n = (int)(sizeof(vals) / sizeof(vals[0]));
	for (int i = 0; i < n && i < cap; ++i)
		out[i] = vals[i];
	return n;
}

int ref_bbox_intersect(const void* bbox24, const float* origin, const float* invDir, const int* dirIsNeg, float lowest) {
	BBox b;
	std::memcpy(&b, bbox24, sizeof(BBox));
	int neg[3] = { dirIsNeg[0], dirIsNeg[1], dirIsNeg[2] };
	return b.intersect(glm::vec3(origin[0], origin[1], origin[2]), glm::vec3(invDir[0], invDir[1], invDir[2]), neg, lowest) ? 1 : 0;
}

float ref_triangle_intersect(const void* tri40, const float* origin, const float* direction) {
	Triangle t;
	std::memcpy(&t, tri40, sizeof(Triangle));
	return t.intersect(glm::vec3(origin[0], origin[1], origin[2]), glm::vec3(direction[0], direction[1], direction[2]));
}

/* host BBox ops: out = {surfaceArea, largestExtent} after adding n vertices to a default BBox; bbox_out gets the 24 bytes */
void ref_bbox_host_ops(const float* vertices, int n, void* bbox_out, float* out2) {
	BBox b;
	for (int i = 0; i < n; ++i)
		b.addVertex(glm::vec3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]));
	std::memcpy(bbox_out, &b, sizeof(BBox));
	out2[0] = b.surfaceArea();
	out2[1] = (float)b.largestExtent();
}

/* closest hit over a batch of RayQueue records (60 B each, updated in place) */
void ref_bvh_intersect(const void* nodes, const void* prims, void* rays, int n, int* hit_out, int* traversals_out) {
	CachedBVH bvh;
	bvh.nodes = (BVH::BVHNode*)nodes;
	bvh.primitives = (Triangle*)prims;
	RayQueue* r = (RayQueue*)rays;
	for (int i = 0; i < n; ++i) {
		if (traversals_out) {
			RayQueue copy = r[i];
			int trav = 0;
			bvh.intersect_debug(copy, &trav);
			traversals_out[i] = trav;
		}
		hit_out[i] = bvh.intersect(r[i]) ? 1 : 0;
	}
}

/* any hit over a batch of ShadowQueue records (44 B each) */
void ref_bvh_intersect_simple(const void* nodes, const void* prims, void* rays, int n, int* hit_out) {
	CachedBVH bvh;
	bvh.nodes = (BVH::BVHNode*)nodes;
	bvh.primitives = (Triangle*)prims;
	ShadowQueue* r = (ShadowQueue*)rays;
	for (int i = 0; i < n; ++i) {
		hit_out[i] = bvh.intersectSimple(r[i], r[i].closestDistance) ? 1 : 0;
	}
}


/* ---- the vendored glm itself (Dependencies/glm-0.9.9.3): the vector functions the path calls ------------------
 * kernel.cu / sunsky.cu / bvh.cpp go through glm::dot, cross, normalize, length, reflect, min, max, clamp, mix,
 * smoothstep, pow, exp and the vec3 operators; the restatement (oracle/orc_internal.h v3*) and the device code
 * (hip/vecmath.hpp) re-implement them in glm's evaluation order.  This export runs the REAL glm on arrays so that both
 * can be pinned to it bit for bit (tests/golden/ref_glm.npz).  n elements; a, b, c are float3 arrays, out is float3.
 * op: 0 dot  1 cross  2 normalize  3 length  4 reflect  5 min  6 max  7 clamp(a, b.x, b.y)  8 mix(a, b, c.x)
 *     9 smoothstep(c.x, c.y, a.x)  10 pow(a, b)  11 a / c.x  12 a * c.x  13 c.x * a  14 exp(a)  15 a * b  16 a / b  17 -a
 *     18 a + b  19 a - b  */
int ref_glm(int op, const float* a, const float* b, const float* c, int n, float* out) {
	for (int i = 0; i < n; ++i) {
		const glm::vec3 A(a[3 * i], a[3 * i + 1], a[3 * i + 2]), B(b[3 * i], b[3 * i + 1], b[3 * i + 2]), Cc(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
		glm::vec3 r(0.0f);
		switch (op) {
		case 0: r.x = glm::dot(A, B); break;
		case 1: r = glm::cross(A, B); break;
		case 2: r = glm::normalize(A); break;
		case 3: r.x = glm::length(A); break;
		case 4: r = glm::reflect(A, B); break;
		case 5: r = glm::min(A, B); break;
		case 6: r = glm::max(A, B); break;
		case 7: r = glm::clamp(A, B.x, B.y); break;
		case 8: r = glm::mix(A, B, Cc.x); break;
		case 9: r.x = glm::smoothstep(Cc.x, Cc.y, A.x); break;
		case 10: r = glm::pow(A, B); break;
		case 11: r = A / Cc.x; break;
		case 12: r = A * Cc.x; break;
		case 13: r = Cc.x * A; break;
		case 14: r = glm::exp(A); break;
		case 15: r = A * B; break;
		case 16: r = A / B; break;
		case 17: r = -A; break;
		case 18: r = A + B; break;
		case 19: r = A - B; break;
		default: return -1;
		}
		out[3 * i] = r.x;
		out[3 * i + 1] = r.y;
		out[3 * i + 2] = r.z;
	}
	return 0;
}

} /* extern "C" */
