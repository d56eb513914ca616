// host.hpp -- host-side pieces of libtyrant_hip.so (C++17, compiled by hipcc as host code)
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "../../../include/tyr_c.h"
#include "device_build.hpp"
#include "../hip/kernels.hpp"
#include "../hip/sunsky.hpp"
#include "../hip/traverse.hpp"
#include "../hip/vecmath.hpp"

namespace tyr {

// host/sun_setup.cpp
void sun_setup(float sun_x, float sun_y, SunParams& out);

// host/bvh_build.cpp -- class BVH of the reference (bvh.h:49-108, bvh.cpp:3-225)
int bvh_build(tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, int32_t algo);
void triangle_bboxes(const tyr_triangle* prims, int32_t n, tyr_bbox* out);
void set_build_threads(int threads); // 0 = automatic (TYR_BUILD_THREADS or min(16, cores))

int build_threads();                 // what the setting resolves to right now (>= 1)
// hip/bvh_build_dev.hip -- the same build (SAH) on the device, the same bytes; host arrays in and out
int bvh_build_device(int device, tyr_triangle* prims, int32_t n, const tyr_bbox* bboxes, tyr_bvh_node* nodes_out, double* seconds_out);
// hip/bvh_layout_dev.hip -- build_device_layout's quad records and triangles made ON the device from device copies of the
// reference's arrays, the same bytes.  On TYR_OK out.quads / out.tris are hipMalloc'ed and the caller's; TYR_ERR_UNSUPPORTED =
// a tree the host pass has to do (pair nodes wanted, over-long leaves, one-leaf trees, malformed input: its error codes stay the host's)
struct DeviceTreeLayout {
	float4* quads = nullptr;
	float4* tris = nullptr;
	uint32_t nQuads = 0, nStaged = 0, quadMaxStack = 0, quadRootRef = 0;
	float rootMin[3] = { 0.f, 0.f, 0.f }, rootMax[3] = { 0.f, 0.f, 0.f };
};
int layout_on_device(const tyr_bvh_node* dNodes, int32_t nNodes, const tyr_triangle* dPrims, int32_t nPrims, DeviceTreeLayout& out, hipStream_t stream);

// host/bvh_layout.cpp -- flat reference nodes -> device quad nodes (+ pair nodes for the counting build) + 48-byte triangles
// an array of floats that is NOT zeroed when it is sized: every element is written by the layout's (parallel) passes, and a
// serial memset of C5's 0.9 GB would cost as much as those passes together
struct FloatBuf {
	std::unique_ptr<float[]> p;
	size_t n = 0;
	void resize(size_t k) {
		p.reset(k ? new float[k] : nullptr);
		n = k;
	}
	void clear() { resize(0); }
	float* data() { return p.get(); }
	const float* data() const { return p.get(); }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	float& operator[](size_t i) { return p[i]; }
	const float& operator[](size_t i) const { return p[i]; }
};
struct DeviceLayout {
	FloatBuf pairNodes; // 16 floats per pair node (only when asked for)
	FloatBuf quadNodes; // 32 floats per quad node (hip/traverse.hpp "QuadNode")
	uint32_t quadRootRef = 0;
	uint32_t nQuads = 0;
	uint32_t nStaged = 0; // leading records that are the top of the tree in breadth-first order
	uint32_t quadMaxStack = 0; // the most entries a traversal of the quad tree can ever have on its stack (whatever the ray)
	FloatBuf tris;      // 12 floats per triangle
	float rootMin[3], rootMax[3];
	uint32_t rootRef;
	uint32_t nPairs;
};
// returns TYR_OK or TYR_ERR_INVALID (malformed tree -- the array must be the reference's depth-first layout: first child = index
// + 1, every subtree a contiguous range, bvh.cpp:195-202 --, non-finite geometry, too many primitives).  wantPairs: also lay out
// the pair nodes (the counting build, BVH_DEBUG); rootRef and nPairs = 0 otherwise.  Runs on build_threads() threads; the bytes
// do not depend on their number.
int build_device_layout(const tyr_bvh_node* nodes, int32_t nNodes, const tyr_triangle* prims, int32_t nPrims, DeviceLayout& out, bool wantPairs = true);

} // namespace tyr
