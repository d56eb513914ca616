// tyrant/Bbox.h -- struct BBox and its host operations (Bbox.h:3-36, Bbox.cpp:3-14).
// BBox::intersect (Bbox.h:38-62) runs on the device (hip/traverse.hpp slab_test).
#pragma once
#include "variables.h"
namespace tyrant {
struct BBox {
	vec3 bounds[2] = { { 1e10f, 1e10f, 1e10f }, { -1e10f, -1e10f, -1e10f } }; // Bbox.h:5
	// glibc fmin/fmax: the first argument wins ties
	static float lo(float a, float b) { return (b < a) ? b : a; }
	static float hi(float a, float b) { return (b > a) ? b : a; }
	BBox& addVertex(const vec3& v) { // Bbox.h:8-14
		bounds[0] = { lo(bounds[0].x, v.x), lo(bounds[0].y, v.y), lo(bounds[0].z, v.z) };
		bounds[1] = { hi(bounds[1].x, v.x), hi(bounds[1].y, v.y), hi(bounds[1].z, v.z) };
		return *this;
	}
	vec3 diagonal() const { return { bounds[1].x - bounds[0].x, bounds[1].y - bounds[0].y, bounds[1].z - bounds[0].z }; }
	float surfaceArea() const { // Bbox.h:18-21
		const vec3 d = diagonal();
		return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
	}
	float volume() const {
		const vec3 d = diagonal();
		return d.x * d.y * d.z;
	}
	int largestExtent() const { // Bbox.h:28-36
		const vec3 d = diagonal();
		if (d.x > d.y && d.x > d.z)
			return 0;
		return (d.y > d.z) ? 1 : 2;
	}
};
static_assert(sizeof(BBox) == 24, "BBox layout");
inline BBox Union(const BBox& a, const BBox& b) { // Bbox.cpp:3-14
	BBox r;
	r.bounds[0] = { BBox::lo(a.bounds[0].x, b.bounds[0].x), BBox::lo(a.bounds[0].y, b.bounds[0].y), BBox::lo(a.bounds[0].z, b.bounds[0].z) };
	r.bounds[1] = { BBox::hi(a.bounds[1].x, b.bounds[1].x), BBox::hi(a.bounds[1].y, b.bounds[1].y), BBox::hi(a.bounds[1].z, b.bounds[1].z) };
	return r;
}
} // namespace tyrant
